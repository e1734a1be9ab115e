// Kernels of the FSMN voice-activity network on gfx950: the conversion of the features to the GEMM's operand type, the
// causal memory block with its per-stream cache, the softmax over the output classes and the per-frame level.  The
// network's matrix products run on the encoder's MFMA GEMM (gemm.hip).
#include "common.h"
#include "elem.h"
#include "vad.h"

namespace asrfe {

namespace {
__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 32));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}
template <bool F16>
__device__ __forceinline__ float from_elem(uint16_t u) {
    if constexpr (F16) return static_cast<float>(__builtin_bit_cast(_Float16, u));
    else return __uint_as_float(static_cast<uint32_t>(u) << 16);
}
constexpr int kFsmnRows = 32;  // output rows per workgroup of the memory block
}  // namespace

// ---- features f32 [M][K] -> 16-bit [M][ld], zero in the columns K .. ld - 1 (K and ld multiples of 8) -------------
template <bool F16>
__global__ __launch_bounds__(256) void vad_cast_kernel(const float* __restrict__ in, int64_t M, int K, uint16_t* __restrict__ out, int ld) {
    saturate_fp16_conversions<F16>();
    const int groups = ld >> 3;
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t row = idx / groups;
    const int g = static_cast<int>(idx - row * groups);
    if (row >= M) return;
    u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (8 * g < K) {
        const float4* src = reinterpret_cast<const float4*>(in + row * K + 8 * g);
        const float4 a = src[0], c = src[1];
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = to_elem<F16>(v[e]);
    }
    *reinterpret_cast<u16x8*>(out + row * ld + 8 * g) = o;
}

// ---- memory block: q[t][c] = p[t][c] + sum_j taps[j][c] * p[t - (lorder - 1) + j][c], rows before the call from the cache --
// One workgroup = kFsmnRows rows of one stream; the rows it reads (its own and the lorder - 1 before them) are staged in
// LDS as floats, then a thread walks one channel of one row.  Consecutive threads own consecutive channels: whole lines.
template <bool F16>
__global__ __launch_bounds__(256) void vad_fsmn_kernel(const VadFsmnParams p) {
    extern __shared__ float rows[];  // [kFsmnRows + hist][C]
    saturate_fp16_conversions<F16>();
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * kFsmnRows;
    const int hist = p.lorder - 1;
    const int C = p.C;
    const int64_t base = static_cast<int64_t>(b) * p.T;
    const float* cache = p.cache ? p.cache + static_cast<int64_t>(b) * p.cache_stride : nullptr;
    for (int i = threadIdx.x; i < (kFsmnRows + hist) * C; i += 256) {
        const int r = i / C, c = i - r * C;
        const int t = t0 - hist + r;
        float v = 0.0f;
        if (t < 0) v = cache ? cache[(hist + t) * C + c] : 0.0f;
        else if (t < p.T) v = from_elem<F16>(p.p[(base + t) * p.ldp + c]);
        rows[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kFsmnRows * p.ldq; i += 256) {
        const int r = i / p.ldq, c = i - r * p.ldq;
        const int t = t0 + r;
        if (t >= p.T) break;
        float acc = 0.0f;
        if (c < C) {
            acc = rows[(r + hist) * C + c];
            for (int j = 0; j < p.lorder; ++j) acc = fmaf(p.taps[j * C + c], rows[(r + j) * C + c], acc);
        }
        p.q[(base + t) * p.ldq + c] = to_elem<F16>(acc);
    }
}

// ---- cache[b] <- the last hist rows of (cache[b] ++ p[b][0 .. len)) ------------------------------------------------
// One workgroup per stream; everything is read before anything is written (a short chunk moves old rows forward).
template <bool F16>
__global__ __launch_bounds__(256) void vad_cache_update_kernel(const uint16_t* __restrict__ p, int ldp, const int32_t* __restrict__ lens,
                                                               int T, int C, int hist, float* __restrict__ cache_all,
                                                               int64_t cache_stride) {
    const int b = blockIdx.x;
    int len = lens ? lens[b] : T;
    len = len < 0 ? 0 : (len > T ? T : len);
    if (len == 0) return;  // block-uniform
    float* cache = cache_all + static_cast<int64_t>(b) * cache_stride;
    const int n = hist * C;
    float v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int i = threadIdx.x + 256 * k;
        v[k] = 0.0f;
        if (i < n) {
            const int r = i / C, c = i - r * C;
            const int t = len - hist + r;
            v[k] = t >= 0 ? from_elem<F16>(p[(static_cast<int64_t>(b) * T + t) * ldp + c]) : cache[(hist + t) * C + c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < n) cache[i] = v[k];
    }
}

// ---- softmax over the first n columns of a row; one wave per row ---------------------------------------------------
__global__ __launch_bounds__(256) void vad_softmax_kernel(const float* __restrict__ logits, int ld, int64_t M, int n,
                                                          float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* src = logits + row * ld;
    float v[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < n ? src[c] : -3.0e38f;
        mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = lane + 64 * k;
        v[k] = c < n ? __expf(v[k] - mx) : 0.0f;
        sum += v[k];
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = lane + 64 * k;
        if (c < n) scores[row * n + c] = v[k] * inv;
    }
}

// ---- level of a frame: 10 log10(sum over its 400 samples of x^2 + 1e-6), x in [-1, 1]; one wave per frame --------------
template <typename PcmT>
__device__ __forceinline__ float unit_sample(PcmT v);
template <>
__device__ __forceinline__ float unit_sample<int16_t>(int16_t v) { return static_cast<float>(v) * (1.0f / 32768.0f); }
template <>
__device__ __forceinline__ float unit_sample<float>(float v) { return v; }

template <typename PcmT, bool WIRE>
__global__ __launch_bounds__(256) void frame_decibel_kernel(const PcmT* __restrict__ pcm, int64_t stride, const int32_t* __restrict__ lens,
                                                            int t_max, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= t_max) return;
    const int64_t len = lens[b];
    const int64_t n_frames = len < kFrameLen ? 0 : 1 + (len - kFrameLen) / kFrameShift;
    float level = 0.0f;
    if (f < n_frames) {
        const PcmT* src = pcm + static_cast<int64_t>(b) * stride + static_cast<int64_t>(kFrameShift) * f;
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int i = lane + 64 * k;
            if (i < kFrameLen) {
                float x = unit_sample<PcmT>(src[i]);
                // a session float sample takes the int16 hand-off first, as the frontend kernels do
                if constexpr (WIRE) x = __builtin_truncf(__builtin_fminf(__builtin_fmaxf(x, -1.0f), 1.0f) * 32767.0f) * (1.0f / 32768.0f);
                ss = fmaf(x, x, ss);
            }
        }
        ss = wave_sum(ss);
        level = 10.0f * log10f(ss + 1.0e-6f);
    }
    if (lane == 0) out[static_cast<int64_t>(b) * t_max + f] = level;
}

// ---- launchers ------------------------------------------------------------------------------------------------------
namespace {
int32_t after_launch(asrfe_ctx* ctx, const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return ASRFE_OK;
}
}  // namespace

int32_t launch_vad_cast(asrfe_ctx* ctx, const float* in, int64_t M, int K, uint16_t* out, int ld, hipStream_t s) {
    if (M <= 0) return ASRFE_OK;
    if (K % 8 != 0 || ld % 8 != 0 || ld < K) return set_err(ctx, ASRFE_ERR_SHAPE, "vad cast: widths must be multiples of 8");
    const int64_t n = M * (ld / 8);
    const dim3 grid(static_cast<unsigned>((n + 255) / 256)), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(vad_cast_kernel<true>, grid, block, 0, s, in, M, K, out, ld);
    else hipLaunchKernelGGL(vad_cast_kernel<false>, grid, block, 0, s, in, M, K, out, ld);
    return after_launch(ctx, "vad cast launch");
}

int32_t launch_vad_fsmn(asrfe_ctx* ctx, const VadFsmnParams& p, int B, hipStream_t s) {
    if (B <= 0 || p.T <= 0) return ASRFE_OK;
    const size_t lds = static_cast<size_t>(kFsmnRows + p.lorder - 1) * p.C * sizeof(float);
    if (p.lorder < 1 || p.lorder > 33 || p.C < 1 || p.C > p.ldq || p.C > p.ldp || lds > 64 * 1024)
        return set_err(ctx, ASRFE_ERR_SHAPE, "vad fsmn: lorder in 1..33 and (31 + lorder) * proj_dim floats must fit 64 KiB of LDS");
    const dim3 grid((p.T + kFsmnRows - 1) / kFsmnRows, B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(vad_fsmn_kernel<true>, grid, block, lds, s, p);
    else hipLaunchKernelGGL(vad_fsmn_kernel<false>, grid, block, lds, s, p);
    return after_launch(ctx, "vad fsmn launch");
}

int32_t launch_vad_cache_update(asrfe_ctx* ctx, const uint16_t* p, int ldp, const int32_t* lens, int B, int T, int C, int hist,
                                float* cache, int64_t cache_stride, hipStream_t s) {
    if (B <= 0 || T <= 0 || hist <= 0) return ASRFE_OK;
    if (hist * C > 32 * 256) return set_err(ctx, ASRFE_ERR_SHAPE, "vad cache: (lorder - 1) * proj_dim must not exceed 8192");
    const dim3 grid(B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16)
        hipLaunchKernelGGL(vad_cache_update_kernel<true>, grid, block, 0, s, p, ldp, lens, T, C, hist, cache, cache_stride);
    else
        hipLaunchKernelGGL(vad_cache_update_kernel<false>, grid, block, 0, s, p, ldp, lens, T, C, hist, cache, cache_stride);
    return after_launch(ctx, "vad cache launch");
}

int32_t launch_vad_softmax(asrfe_ctx* ctx, const float* logits, int ld, int64_t M, int n, float* scores, hipStream_t s) {
    if (M <= 0) return ASRFE_OK;
    if (n < 1 || n > 1024 || n > ld) return set_err(ctx, ASRFE_ERR_SHAPE, "vad softmax: 1..1024 classes");
    const dim3 grid(static_cast<unsigned>((M + 3) / 4)), block(256);
    hipLaunchKernelGGL(vad_softmax_kernel, grid, block, 0, s, logits, ld, M, n, scores);
    return after_launch(ctx, "vad softmax launch");
}

int32_t launch_frame_decibel(asrfe_ctx* ctx, const void* pcm, int pcm_dtype, int64_t stride, const int32_t* lens, int B, int t_max,
                             float* out, hipStream_t s) {
    if (B <= 0 || t_max <= 0) return ASRFE_OK;
    const dim3 grid((t_max + 3) / 4, B), block(256);
    switch (pcm_dtype) {
        case ASRFE_PCM_I16:
            hipLaunchKernelGGL((frame_decibel_kernel<int16_t, false>), grid, block, 0, s, static_cast<const int16_t*>(pcm), stride, lens, t_max, out);
            break;
        case ASRFE_PCM_F32:
            hipLaunchKernelGGL((frame_decibel_kernel<float, false>), grid, block, 0, s, static_cast<const float*>(pcm), stride, lens, t_max, out);
            break;
        case ASRFE_PCM_F32_WIRE:
            hipLaunchKernelGGL((frame_decibel_kernel<float, true>), grid, block, 0, s, static_cast<const float*>(pcm), stride, lens, t_max, out);
            break;
        default: return set_err(ctx, ASRFE_ERR_BAD_ARG, "frame decibel: bad pcm_dtype");
    }
    return after_launch(ctx, "frame decibel launch");
}

}  // namespace asrfe
