// HiFi-GAN generator kernels on gfx950: every convolution (plain, dilated, transposed) as an implicit GEMM on the
// matrix cores, plus the memory-bound pieces around them (input layout change, stage average, output convolution).
#include "common.h"
#include "elem.h"
#include "vocoder.h"

namespace asrfe {

namespace {
template <bool F16>
__device__ __forceinline__ float from_elem(uint16_t u) {
    if constexpr (F16) return static_cast<float>(__builtin_bit_cast(_Float16, u));
    else return __uint_as_float(static_cast<uint32_t>(u) << 16);
}
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }
}  // namespace

// One wave = 32 output channels x 32 positions; a workgroup = 4 waves = 128 consecutive positions of one channel tile,
// one batch item and one phase.  Both operands come straight from global memory: the weight rows of a channel tile are
// shared by every position tile (L2), the activation rows by the taps that overlap them (L1).  MFMA operand layout:
// lane (l31, h) holds 8 consecutive k = 8 consecutive input channels of one tap, of weight row l31 (A) and of position
// l31 (B); accumulator register 4 g + e is output channel 8 g + 4 h + e of position l31.
template <bool F16>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    saturate_fp16_conversions<F16>();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / p.phases, ph = blockIdx.z - b * p.phases;
    const int co0 = blockIdx.y * 32;
    const int q = blockIdx.x * 128 + wave * 32 + l31;
    const int K = p.taps * p.Ci;
    const uint16_t* wrow = p.w + (static_cast<int64_t>(ph) * p.Co + co0 + l31) * K + 8 * h;
    const uint16_t* xb = p.x + static_cast<int64_t>(b) * p.Tin * p.Ci + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int nblk = p.Ci >> 4;
    for (int tap = 0; tap < p.taps; ++tap) {
        const int ti = q + p.off0 + tap * p.dtap;
        const bool ok = ti >= 0 && ti < p.Tin;
        const uint16_t* xr = xb + static_cast<int64_t>(ok ? ti : 0) * p.Ci;
        const uint16_t* wr = wrow + tap * p.Ci;
        for (int cb = 0; cb < nblk; ++cb) {
            u16x8 xf = *reinterpret_cast<const u16x8*>(xr + 16 * cb);
            if (!ok) xf = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
            const u16x8 wf = *reinterpret_cast<const u16x8*>(wr + 16 * cb);
            acc = mfma32<F16>(wf, xf, acc);
        }
    }
    const int to = q * p.out_mul + p.out_add0 + ph * p.out_add_step;
    if (q >= p.q_count || to < 0 || to >= p.Tout) return;
    const int64_t orow = (static_cast<int64_t>(b) * p.Tout + to) * p.Co + co0 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t o = orow + 8 * g;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co0 + 8 * g + 4 * h);
        float v0 = acc[4 * g] + b4.x, v1 = acc[4 * g + 1] + b4.y, v2 = acc[4 * g + 2] + b4.z, v3 = acc[4 * g + 3] + b4.w;
        if (p.res_in) {
            const float4 r = *reinterpret_cast<const float4*>(p.res_in + o);
            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
        }
        if (p.res_out) *reinterpret_cast<float4*>(p.res_out + o) = make_float4(v0, v1, v2, v3);
        if (p.sum_mode == 1) {
            *reinterpret_cast<float4*>(p.sum_out + o) = make_float4(v0, v1, v2, v3);
        } else if (p.sum_mode == 2) {
            const float4 t = *reinterpret_cast<const float4*>(p.sum_out + o);
            *reinterpret_cast<float4*>(p.sum_out + o) = make_float4(t.x + v0, t.y + v1, t.z + v2, t.w + v3);
        }
        if (p.y16)
            *reinterpret_cast<uint2*>(p.y16 + o) = make_uint2(pack_elem2<F16>(lrelu(v0, p.slope), lrelu(v1, p.slope)),
                                                              pack_elem2<F16>(lrelu(v2, p.slope), lrelu(v3, p.slope)));
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void mel_to_ntc_kernel(const float* __restrict__ mel, int n_mels, int T, uint16_t* __restrict__ out) {
    saturate_fp16_conversions<F16>();
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // over T * n_mels, channel fastest
    if (i >= T * n_mels) return;
    const int t = i / n_mels, c = i - t * n_mels;
    out[static_cast<int64_t>(b) * T * n_mels + i] = to_elem<F16>(mel[(static_cast<int64_t>(b) * n_mels + c) * T + t]);
}

template <bool F16>
__global__ __launch_bounds__(256) void scale_act_kernel(const float* __restrict__ sum, float scale, float slope, int64_t n4,
                                                        uint16_t* __restrict__ y16) {
    saturate_fp16_conversions<F16>();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(sum)[i];
    reinterpret_cast<uint2*>(y16)[i] = make_uint2(pack_elem2<F16>(lrelu(v.x * scale, slope), lrelu(v.y * scale, slope)),
                                                  pack_elem2<F16>(lrelu(v.z * scale, slope), lrelu(v.w * scale, slope)));
}

template <bool F16>
__global__ __launch_bounds__(256) void conv_post_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int T, int C, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc = bias[0];
    for (int tap = 0; tap < 7; ++tap) {
        const int ti = t + tap - 3;
        if (ti < 0 || ti >= T) continue;
        const u16x8* xr = reinterpret_cast<const u16x8*>(x + (static_cast<int64_t>(b) * T + ti) * C);
        const float* wr = w + tap * C;
        for (int c8 = 0; c8 < (C >> 3); ++c8) {
            const u16x8 v = xr[c8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(wr[8 * c8 + e], from_elem<F16>(v[e]), acc);
        }
    }
    wav[static_cast<int64_t>(b) * T + t] = tanhf(acc);
}

#define VOC_CHECK_LAUNCH(who)                                                                                         \
    do {                                                                                                              \
        const hipError_t e__ = hipGetLastError();                                                                     \
        if (e__ != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string(who " launch: ") + hipGetErrorString(e__)); \
    } while (0)

int32_t launch_conv_mfma(asrfe_ctx* ctx, const ConvParams& p, hipStream_t s) {
    if (p.B <= 0 || p.q_count <= 0) return ASRFE_OK;
    if (p.Ci % 16 != 0 || p.Co % 32 != 0 || p.taps < 1 || p.phases < 1 || !p.x || !p.w || !p.bias)
        return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: input channels must be a multiple of 16, output channels of 32");
    if (static_cast<int64_t>(p.B) * p.phases > 65535) return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: batch * stride too large");
    const dim3 grid((p.q_count + 127) / 128, p.Co / 32, p.B * p.phases), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(conv_mfma_kernel<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(conv_mfma_kernel<false>, grid, block, 0, s, p);
    VOC_CHECK_LAUNCH("vocoder convolution");
    return ASRFE_OK;
}

int32_t launch_mel_to_ntc(asrfe_ctx* ctx, const float* mel, int B, int n_mels, int T, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return ASRFE_OK;
    const dim3 grid((T * n_mels + 255) / 256, B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(mel_to_ntc_kernel<true>, grid, block, 0, s, mel, n_mels, T, out);
    else hipLaunchKernelGGL(mel_to_ntc_kernel<false>, grid, block, 0, s, mel, n_mels, T, out);
    VOC_CHECK_LAUNCH("mel layout");
    return ASRFE_OK;
}

int32_t launch_scale_act(asrfe_ctx* ctx, const float* sum, float scale, float slope, int64_t n, uint16_t* y16, hipStream_t s) {
    if (n <= 0) return ASRFE_OK;
    const int64_t n4 = n / 4;  // (channel counts are multiples of 32)
    const dim3 grid(static_cast<unsigned>((n4 + 255) / 256)), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(scale_act_kernel<true>, grid, block, 0, s, sum, scale, slope, n4, y16);
    else hipLaunchKernelGGL(scale_act_kernel<false>, grid, block, 0, s, sum, scale, slope, n4, y16);
    VOC_CHECK_LAUNCH("stage average");
    return ASRFE_OK;
}

int32_t launch_conv_post(asrfe_ctx* ctx, const uint16_t* x, const float* w, const float* bias, int B, int T, int C, float* wav,
                         hipStream_t s) {
    if (B <= 0 || T <= 0) return ASRFE_OK;
    const dim3 grid((T + 255) / 256, B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(conv_post_kernel<true>, grid, block, 0, s, x, w, bias, T, C, wav);
    else hipLaunchKernelGGL(conv_post_kernel<false>, grid, block, 0, s, x, w, bias, T, C, wav);
    VOC_CHECK_LAUNCH("output convolution");
    return ASRFE_OK;
}

}  // namespace asrfe
