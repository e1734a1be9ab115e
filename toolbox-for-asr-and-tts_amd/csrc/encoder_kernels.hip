// Memory-bound kernels of the SAN-M encoder on gfx950: sinusoidal position table and the LayerNorm
// variants (one wave per row, fp32 statistics, 16-byte loads).
#include "common.h"
#include "elem.h"
#include "encoder.h"

namespace asrfe {

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

__global__ __launch_bounds__(256) void pe_kernel(const float* __restrict__ inv, int Tp, int D, float* __restrict__ pe) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= D) return;
    const int half = D / 2;
    const float pos = static_cast<float>(t + 1);
    const float a = pos * inv[i < half ? i : i - half];
    pe[static_cast<int64_t>(t) * D + i] = i < half ? sinf(a) : cosf(a);
}

struct LnParams {
    const float* x;       // [M][D]
    const float* pe;      // [Tp][D] (input mode)
    float xscale;
    const float* g;
    const float* b;
    float eps;
    int M, Tp, D;
    void* out;
    int ldo;
    const int32_t* lens;  // final mode
};

// MODE 0: x -> 16-bit h.  MODE 1: (x * xscale + pe) -> 16-bit h with zero pad columns.
// MODE 2: final, f32 out, masked.  MODE 3: final, 16-bit out, masked.  F16 picks fp16 over bf16.
template <int NV, int MODE, bool F16>
__global__ __launch_bounds__(256) void ln_kernel(const LnParams p) {
    saturate_fp16_conversions<F16>();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const int nvec = p.D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(p.x + static_cast<int64_t>(row) * p.D);
    const int t = row % p.Tp;
    float4 v[NV];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) {
            v[k] = xr[idx];
            if constexpr (MODE == 1) {
                const float4 q = reinterpret_cast<const float4*>(p.pe + static_cast<int64_t>(t) * p.D)[idx];
                v[k].x = v[k].x * p.xscale + q.x;
                v[k].y = v[k].y * p.xscale + q.y;
                v[k].z = v[k].z * p.xscale + q.z;
                v[k].w = v[k].w * p.xscale + q.w;
            }
            sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        } else {
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(sum) / static_cast<float>(p.D);
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        if (idx < nvec) {
            const float a = v[k].x - mean, b = v[k].y - mean, c = v[k].z - mean, d = v[k].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = wave_sum(sq) / static_cast<float>(p.D);
    const float rstd = 1.0f / sqrtf(var + p.eps);
    bool live = true;
    if constexpr (MODE >= 2) live = t < p.lens[row / p.Tp];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int idx = lane + 64 * k;
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nvec && live) {
            float4 g = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.g) {  // null: the scale and shift are folded into the linear layer that follows
                g = reinterpret_cast<const float4*>(p.g)[idx];
                bb = reinterpret_cast<const float4*>(p.b)[idx];
            }
            y.x = (v[k].x - mean) * rstd * g.x + bb.x;
            y.y = (v[k].y - mean) * rstd * g.y + bb.y;
            y.z = (v[k].z - mean) * rstd * g.z + bb.z;
            y.w = (v[k].w - mean) * rstd * g.w + bb.w;
        }
        if (idx < (p.ldo >> 2)) {
            if constexpr (MODE == 2) {
                reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + static_cast<int64_t>(row) * p.ldo)[idx] = y;
            } else {
                reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + static_cast<int64_t>(row) * p.ldo)[idx] =
                    make_uint2(pack_elem2<F16>(y.x, y.y), pack_elem2<F16>(y.z, y.w));
            }
        }
    }
}

template <int NV, int MODE>
static int32_t launch_ln(asrfe_ctx* ctx, const LnParams& p, bool f16, hipStream_t s, const char* who) {
    if (p.M <= 0) return ASRFE_OK;
    if (p.D % 4 != 0 || p.ldo % 4 != 0 || p.D > NV * 256 || p.ldo > NV * 256 || p.ldo < p.D)
        return set_err(ctx, ASRFE_ERR_SHAPE, std::string(who) + ": unsupported row width");
    if (f16) hipLaunchKernelGGL((ln_kernel<NV, MODE, true>), dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((ln_kernel<NV, MODE, false>), dim3((p.M + 3) / 4), dim3(256), 0, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string(who) + " launch: " + hipGetErrorString(e));
    return ASRFE_OK;
}

int32_t launch_pe_table(asrfe_ctx* ctx, const float* inv, int Tp, int D, float* pe, hipStream_t s) {
    if (Tp <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(pe_kernel, dim3((D + 255) / 256, Tp), dim3(256), 0, s, inv, Tp, D, pe);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("pe launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

int32_t launch_ln_input(asrfe_ctx* ctx, const float* feats, const float* pe, float xscale, const float* g,
                        const float* bta, float eps, int M, int Tp, int D, uint16_t* hout, int ldh, hipStream_t s) {
    LnParams p{feats, pe, xscale, g, bta, eps, M, Tp, D, hout, ldh, nullptr};
    return launch_ln<3, 1>(ctx, p, ctx->cfg.enc_compute_dtype == ASRFE_DT_F16, s, "ln_input");
}

int32_t launch_ln512(asrfe_ctx* ctx, const float* x, const float* g, const float* bta, float eps, int M, uint16_t* hout,
                     hipStream_t s) {
    LnParams p{x, nullptr, 1.0f, g, bta, eps, M, 1, 512, hout, 512, nullptr};
    return launch_ln<2, 0>(ctx, p, ctx->cfg.enc_compute_dtype == ASRFE_DT_F16, s, "ln512");
}

int32_t launch_ln_final(asrfe_ctx* ctx, const float* x, const float* g, const float* bta, float eps, int B, int Tp,
                        const int32_t* lens, void* out, int out_dtype, hipStream_t s) {
    LnParams p{x, nullptr, 1.0f, g, bta, eps, B * Tp, Tp, 512, out, 512, lens};
    if (out_dtype == ASRFE_DT_BF16) return launch_ln<2, 3>(ctx, p, false, s, "ln_final");
    if (out_dtype == ASRFE_DT_F16) return launch_ln<2, 3>(ctx, p, true, s, "ln_final");
    return launch_ln<2, 2>(ctx, p, false, s, "ln_final");
}

}  // namespace asrfe
