// Host-side helpers of the weight-loading calls: named host tensors, shape checks, conversion to the 16-bit
// compute type, LayerNorm folding and uploads that the context owns.
#pragma once

#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace asrfe {

#define ASRFE_HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return set_err(ctx, ASRFE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

struct HostTensor {
    const float* data;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};
typedef std::map<std::string, HostTensor> TensorMap;

// descriptors -> map by name, with `strip` removed from the front of a name that starts with it
inline int32_t tensor_map(asrfe_ctx* ctx, const asrfe_tensor_desc* descs, int32_t n, const char* strip, const char* who,
                          TensorMap& m) {
    const size_t ns = std::strlen(strip);
    for (int i = 0; i < n; ++i) {
        if (!descs[i].name || !descs[i].data_host || descs[i].ndim < 1 || descs[i].ndim > 4)
            return set_err(ctx, ASRFE_ERR_BAD_ARG, std::string(who) + ": bad descriptor");
        HostTensor t;
        t.data = descs[i].data_host;
        t.shape.assign(descs[i].shape, descs[i].shape + descs[i].ndim);
        std::string name = descs[i].name;
        if (ns && name.rfind(strip, 0) == 0) name = name.substr(ns);
        m[name] = t;
    }
    return ASRFE_OK;
}

inline int32_t need(asrfe_ctx* ctx, const TensorMap& m, const std::string& name, std::vector<int64_t> shape,
                    const HostTensor** out, const char* who = "asrfe_load_encoder_weights") {
    auto it = m.find(name);
    if (it == m.end()) return set_err(ctx, ASRFE_ERR_NOT_LOADED, std::string(who) + ": missing tensor " + name);
    if (it->second.shape != shape) return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": bad shape for " + name);
    *out = &it->second;
    return ASRFE_OK;
}

template <typename T>
inline int32_t upload(asrfe_ctx* ctx, const std::vector<T>& host, T** dev) {
    void* p = nullptr;
    ASRFE_HIP_TRY(ctx, hipMalloc(&p, std::max<size_t>(host.size() * sizeof(T), 16)));
    ctx->owned.push_back(p);
    ASRFE_HIP_TRY(ctx, hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T*>(p);
    return ASRFE_OK;
}

inline int32_t up_f32(asrfe_ctx* ctx, const HostTensor* t, float** dev) {
    std::vector<float> v(t->data, t->data + t->numel());
    return upload(ctx, v, dev);
}

inline uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}
inline uint16_t f16_rne(float f) {
    const _Float16 h = static_cast<_Float16>(f);
    uint16_t u;
    std::memcpy(&u, &h, 2);
    return u;
}
inline float f16_to_f32(uint16_t u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return static_cast<float>(h);
}
inline float bf16_to_f32(uint16_t u) {
    const uint32_t w = static_cast<uint32_t>(u) << 16;
    float f;
    std::memcpy(&f, &w, 4);
    return f;
}

// [n][k] float -> 16-bit [n_pad][k_pad] (bf16 or fp16 per the context), zero past n and k
inline int32_t up_16(asrfe_ctx* ctx, const float* w, int64_t n, int64_t k, int64_t n_pad, int64_t k_pad, uint16_t** dev) {
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    std::vector<uint16_t> v(static_cast<size_t>(n_pad) * k_pad, 0);
    for (int64_t r = 0; r < n; ++r)
        for (int64_t c = 0; c < k; ++c) v[r * k_pad + c] = f16 ? f16_rne(w[r * k + c]) : bf16_rne(w[r * k + c]);
    return upload(ctx, v, dev);
}
inline int32_t up_bf16(asrfe_ctx* ctx, const HostTensor* t, int k_pad, uint16_t** dev) {
    return up_16(ctx, t->data, t->shape[0], t->shape[1], t->shape[0], k_pad, dev);
}

// A LayerNorm's scale and shift in front of a linear layer are folded into it:
//     W (g * n + b) + c  =  (W diag(g)) n + (W b + c),        n = the normalised row,
// so the LayerNorm kernels (and the fused feed-forward kernel) emit n without touching g and b.  c may be null
// (a linear layer without bias).
inline void fold_layernorm(const HostTensor* w, const HostTensor* c, const HostTensor* g, const HostTensor* b,
                           std::vector<float>& wf, std::vector<float>& cf) {
    const int64_t n = w->shape[0], k = w->shape[1];
    wf.resize(static_cast<size_t>(n * k));
    cf.resize(static_cast<size_t>(n));
    for (int64_t r = 0; r < n; ++r) {
        double acc = c ? c->data[r] : 0.0;
        for (int64_t i = 0; i < k; ++i) {
            wf[r * k + i] = w->data[r * k + i] * g->data[i];
            acc += static_cast<double>(w->data[r * k + i]) * b->data[i];
        }
        cf[r] = static_cast<float>(acc);
    }
}

// |w . n + c| <= ||w||_2 sqrt(K) + |c| for a normalised row n (||n||_2 = sqrt(K)): the largest such bound over rows
// r0 .. r1 - 1 of a [rows][k] matrix (the fp16 range proof of the load calls)
constexpr double kF16Max = 65504.0;
inline double linear_bound(const std::vector<float>& w, const std::vector<float>& c, int64_t rows, int64_t k, int64_t r0,
                           int64_t r1, std::vector<double>* per_row = nullptr) {
    double worst = 0.0;
    for (int64_t r = r0; r < r1 && r < rows; ++r) {
        double ss = 0.0;
        for (int64_t i = 0; i < k; ++i) ss += static_cast<double>(w[r * k + i]) * w[r * k + i];
        const double b = std::sqrt(ss) * std::sqrt(static_cast<double>(k)) + std::fabs(static_cast<double>(c[r]));
        if (per_row) (*per_row)[r - r0] = b;
        worst = std::fmax(worst, b);
    }
    return worst;
}

}  // namespace asrfe
