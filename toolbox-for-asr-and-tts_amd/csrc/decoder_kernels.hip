// Kernels of the step behind the encoder on gfx950: the CIF predictor's element-wise parts and its integrate-and-fire
// scan, and the decoder's LayerNorm over the feed-forward hidden layer, FSMN memory block and arg-max.  The matrix
// products of both run on the encoder's MFMA GEMM (gemm.hip) and the cross-attention on its attention kernel.
#include "common.h"
#include "decoder.h"
#include "elem.h"

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
template <bool F16>
__device__ __forceinline__ float from_elem(uint16_t u) {
    if constexpr (F16) return static_cast<float>(__builtin_bit_cast(_Float16, u));
    else return __uint_as_float(static_cast<uint32_t>(u) << 16);
}
}  // namespace

// ---- predictor, step 1: the three taps of the convolution side by side, as a 16-bit head and a 16-bit remainder ----
// Row (b, t) of A is [x(t - 1) | x(t) | x(t + 1)] with x = the encoder output inside the utterance and 0 outside
// (rows at or past lens[b] included).  x = hi + lo with hi = x rounded to 16 bits and lo = (x - hi) rounded to 16 bits,
// so that hi W_hi + lo W_hi + hi W_lo reproduces the fp32 product to about 2^-20: the alphas decide WHERE tokens fire,
// and a fire that lands one frame off changes a token's embedding by a whole frame's weight.
// One wave per row; a lane owns 8 consecutive columns of each tap.
template <bool F16>
__global__ __launch_bounds__(256) void pred_split_kernel(const float* __restrict__ enc, const int32_t* __restrict__ lens,
                                                         int M, int Tp, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
    saturate_fp16_conversions<F16>();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = row / Tp, t = row - b * Tp;
    int len = lens[b];
    len = len < 0 ? 0 : (len > Tp ? Tp : len);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int ts = t + j - 1;
        float v[8];
        if (t < len && ts >= 0 && ts < len) {
            const float4* src = reinterpret_cast<const float4*>(enc + (static_cast<int64_t>(b) * Tp + ts) * 512 + lane * 8);
            const float4 a = src[0], c = src[1];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.0f;
        }
        u16x8 h8, l8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h8[e] = to_elem<F16>(v[e]);
            l8[e] = to_elem<F16>(v[e] - from_elem<F16>(h8[e]));
        }
        const int64_t o = static_cast<int64_t>(row) * 1536 + j * 512 + lane * 8;
        *reinterpret_cast<u16x8*>(hi + o) = h8;
        *reinterpret_cast<u16x8*>(lo + o) = l8;
    }
}

// ---- predictor, step 2: alphas = relu(sigmoid(relu(conv) . w + b) * smooth - noise), zero at and past lens[b] ----
__global__ __launch_bounds__(256) void pred_alpha_kernel(const float* __restrict__ conv, const float* __restrict__ w,
                                                         float bias, float smooth, float noise,
                                                         const int32_t* __restrict__ lens, int M, int Tp,
                                                         float* __restrict__ alphas) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = row / Tp, t = row - b * Tp;
    const float4* src = reinterpret_cast<const float4*>(conv + static_cast<int64_t>(row) * 512 + lane * 8);
    const float4* ws = reinterpret_cast<const float4*>(w + lane * 8);
    const float4 a = src[0], c = src[1], wa = ws[0], wc = ws[1];
    float s = fmaxf(a.x, 0.0f) * wa.x + fmaxf(a.y, 0.0f) * wa.y + fmaxf(a.z, 0.0f) * wa.z + fmaxf(a.w, 0.0f) * wa.w +
              fmaxf(c.x, 0.0f) * wc.x + fmaxf(c.y, 0.0f) * wc.y + fmaxf(c.z, 0.0f) * wc.z + fmaxf(c.w, 0.0f) * wc.w;
    s = wave_sum(s) + bias;
    if (lane == 0) {
        const float sg = 1.0f / (1.0f + expf(-s));
        const float al = fmaxf(sg * smooth - noise, 0.0f);
        alphas[row] = t < lens[b] ? al : 0.0f;
    }
}

// ---- predictor, step 3: continuous integrate-and-fire --------------------------------------------------------
// One workgroup per utterance; a thread owns two channels and walks the frames 0 .. len in order (frame len is the
// tail: weight tail_threshold on a zero frame), with the scalar state (integrate, fired or not) recomputed identically
// by every thread.  The arithmetic follows the published loop operation by operation (separate multiply and add), so
// on equal alphas the embeddings agree with an fp32 evaluation to the last bit.
__global__ __launch_bounds__(256) void cif_kernel(const CifParams p) {
    extern __shared__ float sa[];  // the utterance's alphas, tail included
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.Tp ? p.Tp : len);
    for (int t = tid; t <= p.Tp; t += 256) {
        float a = 0.0f;
        if (t < len) a = p.alphas[static_cast<int64_t>(b) * p.Tp + t];
        else if (t == len) a = p.tail_threshold;
        sa[t] = a;
        if (p.alphas_out) p.alphas_out[static_cast<int64_t>(b) * (p.Tp + 1) + t] = a;
    }
    __syncthreads();
    const float2* h = reinterpret_cast<const float2*>(p.enc + static_cast<int64_t>(b) * p.Tp * 512) + tid;
    float2* out = reinterpret_cast<float2*>(p.embeds + static_cast<int64_t>(b) * p.n_max * 512) + tid;
    float integrate = 0.0f;
    float2 frame = make_float2(0.0f, 0.0f);
    double total = 0.0;
    int n = 0;
    const float thr = p.threshold;
    // frames are fetched eight at a time ahead of the scan, which is a chain of dependent scalar steps
    for (int t0 = 0; t0 <= len; t0 += 8) {
        float2 hv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + i;
            hv[i] = t < len ? h[static_cast<int64_t>(t) * 256] : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = t0 + i;
            if (t > len) break;
            const float alpha = sa[t];
            total += static_cast<double>(alpha);
            const float completion = 1.0f - integrate;
            integrate = integrate + alpha;
            const bool fire = integrate >= thr;
            if (fire) integrate = integrate - 1.0f;
            const float cur = fire ? completion : alpha;
            const float remainds = alpha - cur;
            frame.x = __fadd_rn(frame.x, __fmul_rn(cur, hv[i].x));
            frame.y = __fadd_rn(frame.y, __fmul_rn(cur, hv[i].y));
            if (fire) {
                if (n < p.n_max) out[static_cast<int64_t>(n) * 256] = frame;
                ++n;
                frame.x = __fmul_rn(remainds, hv[i].x);
                frame.y = __fmul_rn(remainds, hv[i].y);
            }
        }
    }
    for (int r = n; r < p.n_max; ++r) out[static_cast<int64_t>(r) * 256] = make_float2(0.0f, 0.0f);
    if (tid == 0) {
        // the token count is the floor of the summed weights (the reference's tail process); the fire count above
        // equals it except where the sum sits within rounding of an integer
        int tn = static_cast<int>(floorf(static_cast<float>(total)));
        tn = tn < 0 ? 0 : (tn > p.n_max ? p.n_max : tn);
        p.token_num[b] = tn;
        if (p.n_fires) p.n_fires[b] = n;
    }
}

// ---- the frame at which each token fires: cif_kernel's scalar recurrence on its own, one utterance per wave (lane 0
//      walks the utterance's alphas, tail weight included, as cif_kernel wrote them to alphas_out) ------------------------
__global__ __launch_bounds__(256) void cif_fire_frames_kernel(const float* __restrict__ alphas, int B, int Tp, int n_max,
                                                              float thr, int32_t* __restrict__ fires) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    int32_t* out = fires + static_cast<int64_t>(b) * n_max;
    for (int r = lane; r < n_max; r += 64) out[r] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane != 0) return;
    const float* a = alphas + static_cast<int64_t>(b) * (Tp + 1);
    float integrate = 0.0f;
    int n = 0;
    for (int t = 0; t <= Tp; ++t) {
        integrate = integrate + a[t];
        if (integrate >= thr) {
            integrate = integrate - 1.0f;
            if (n < n_max) out[n] = t;
            ++n;
        }
    }
}

// ---- decoder: LayerNorm over the feed-forward hidden layer (16-bit in, normalised 16-bit out) ------------------
// The scale and shift of this LayerNorm are folded into the second linear layer when the weights are loaded.  One
// wave per row; NV 16-byte pieces per lane (width = 512 NV).
template <int NV, bool F16>
__global__ __launch_bounds__(256) void ln_hidden_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int M,
                                                        int width, float eps) {
    saturate_fp16_conversions<F16>();
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const u16x8* src = reinterpret_cast<const u16x8*>(in + static_cast<int64_t>(row) * width);
    float v[NV][8];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const u16x8 r = src[lane + 64 * k];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[k][e] = from_elem<F16>(r[e]);
            sum += v[k][e];
        }
    }
    const float mean = wave_sum(sum) / static_cast<float>(width);
    float sq = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[k][e] - mean;
            sq += d * d;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / static_cast<float>(width) + eps);
    u16x8* dst = reinterpret_cast<u16x8*>(out + static_cast<int64_t>(row) * width);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        u16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = to_elem<F16>((v[k][e] - mean) * rstd);
        dst[lane + 64 * k] = r;
    }
}

// ---- decoder: x += mask (conv11(mask LN2(t)) + mask LN2(t)) -----------------------------------------------------
// t is the feed-forward output, LN2 the layer's norm2 (scale and shift applied here: a depth-wise convolution with
// zero padding does not commute with a shift).  One workgroup = 16 token rows of one utterance (52 KiB of LDS: three
// workgroups per CU): its four waves normalise the 26 rows the window touches into LDS (rows outside the utterance's
// tokens are zero), then a thread owns two channels and slides the 11 taps down the 16 rows.  With p.h set the
// workgroup also writes the LayerNorm of its 16 updated rows as 16-bit operands (the next product's input; its scale
// and shift live in that product's weights): the rows go back through the LDS, a wave normalises four of them with the
// arithmetic of ln_kernel, in its order, and no separate launch reads the residual stream again.
constexpr int kDecFsmnRows = 16;
template <bool F16>
__global__ __launch_bounds__(256) void dec_fsmn_kernel(const DecFsmnParams p) {
    saturate_fp16_conversions<F16>();
    __shared__ __attribute__((aligned(16))) float tn[(kDecFsmnRows + 10) * 512];
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * kDecFsmnRows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ntok = p.token_num[b];
    ntok = ntok < 0 ? 0 : (ntok > p.N ? p.N : ntok);
    const float4* g4 = reinterpret_cast<const float4*>(p.gamma + lane * 8);
    const float4* b4 = reinterpret_cast<const float4*>(p.beta + lane * 8);
    const float4 ga = g4[0], gb = g4[1], ba = b4[0], bb = b4[1];
    // a wave's rows are all requested before the first one is reduced (one memory round trip, not one per row)
    constexpr int kRowsPerWave = (kDecFsmnRows + 10 + 3) / 4;
    float4 va[kRowsPerWave], vc[kRowsPerWave];
#pragma unroll
    for (int k = 0; k < kRowsPerWave; ++k) {
        const int rr = wave + 4 * k, r = r0 + rr - 5;
        va[k] = vc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (rr < kDecFsmnRows + 10 && r >= 0 && r < ntok) {
            const float4* src = reinterpret_cast<const float4*>(p.t + (static_cast<int64_t>(b) * p.N + r) * 512 + lane * 8);
            va[k] = src[0];
            vc[k] = src[1];
        }
    }
#pragma unroll
    for (int k = 0; k < kRowsPerWave; ++k) {
        const int rr = wave + 4 * k, r = r0 + rr - 5;
        if (rr >= kDecFsmnRows + 10) break;
        float4 oa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ob = oa;
        if (r >= 0 && r < ntok) {
            const float4 a = va[k], c = vc[k];
            const float mean = wave_sum(a.x + a.y + a.z + a.w + c.x + c.y + c.z + c.w) * (1.0f / 512.0f);
            const float dx[8] = {a.x - mean, a.y - mean, a.z - mean, a.w - mean, c.x - mean, c.y - mean, c.z - mean, c.w - mean};
            float sq = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sq += dx[e] * dx[e];
            const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / 512.0f) + p.eps);
            oa = make_float4(dx[0] * rstd * ga.x + ba.x, dx[1] * rstd * ga.y + ba.y, dx[2] * rstd * ga.z + ba.z, dx[3] * rstd * ga.w + ba.w);
            ob = make_float4(dx[4] * rstd * gb.x + bb.x, dx[5] * rstd * gb.y + bb.y, dx[6] * rstd * gb.z + bb.z, dx[7] * rstd * gb.w + bb.w);
        }
        float4* dst = reinterpret_cast<float4*>(tn + rr * 512 + lane * 8);
        dst[0] = oa;
        dst[1] = ob;
    }
    __syncthreads();
    float2 w[11];
#pragma unroll
    for (int j = 0; j < 11; ++j) w[j] = reinterpret_cast<const float2*>(p.taps + j * 512)[tid];
    const float2* col = reinterpret_cast<const float2*>(tn) + tid;
    const int nrow = p.h ? p.N : ntok;  // (the LayerNorm covers the rows past the tokens too, as its own launch did)
    float2 xin[kDecFsmnRows];
#pragma unroll
    for (int i = 0; i < kDecFsmnRows; ++i) {
        const int r = r0 + i;
        xin[i] = make_float2(0.0f, 0.0f);
        if (r < nrow) xin[i] = reinterpret_cast<const float2*>(p.x + (static_cast<int64_t>(b) * p.N + r) * 512)[tid];
    }
#pragma unroll
    for (int i = 0; i < kDecFsmnRows; ++i) {
        const int r = r0 + i;
        if (r >= ntok) break;  // rows past the tokens keep their (finite) residual: nothing reads them as tokens
        float2 acc = col[(i + 5) * 256];
#pragma unroll
        for (int j = 0; j < 11; ++j) {
            const float2 v = col[(i + j) * 256];
            acc.x = fmaf(w[j].x, v.x, acc.x);
            acc.y = fmaf(w[j].y, v.y, acc.y);
        }
        xin[i] = make_float2(xin[i].x + acc.x, xin[i].y + acc.y);
        reinterpret_cast<float2*>(p.x + (static_cast<int64_t>(b) * p.N + r) * 512)[tid] = xin[i];
    }
    if (!p.h) return;
    __syncthreads();  // every thread is done with the normalised rows: the first 16 rows of the region take the new x
#pragma unroll
    for (int i = 0; i < kDecFsmnRows; ++i) reinterpret_cast<float2*>(tn + i * 512)[tid] = xin[i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kDecFsmnRows / 4; ++k) {
        const int i = wave + 4 * k, r = r0 + i;
        if (r >= p.N) break;
        const float4 a = reinterpret_cast<const float4*>(tn + i * 512)[lane], c = reinterpret_cast<const float4*>(tn + i * 512)[lane + 64];
        const float mean = wave_sum(((a.x + a.y) + (a.z + a.w)) + ((c.x + c.y) + (c.z + c.w))) / 512.0f;
        const float a0 = a.x - mean, a1 = a.y - mean, a2 = a.z - mean, a3 = a.w - mean;
        const float c0 = c.x - mean, c1 = c.y - mean, c2 = c.z - mean, c3 = c.w - mean;
        const float var = wave_sum(((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3))) / 512.0f;
        const float rstd = 1.0f / sqrtf(var + p.eps);
        uint2* dst = reinterpret_cast<uint2*>(p.h + (static_cast<int64_t>(b) * p.N + r) * 512);
        dst[lane] = make_uint2(pack_elem2<F16>(a0 * rstd, a1 * rstd), pack_elem2<F16>(a2 * rstd, a3 * rstd));
        dst[lane + 64] = make_uint2(pack_elem2<F16>(c0 * rstd, c1 * rstd), pack_elem2<F16>(c2 * rstd, c3 * rstd));
    }
}

// ---- arg-max over the vocabulary: one wave per row, the lowest index wins a tie; rows at or past the utterance's
//      token count get -1, rows without a maximum (all NaN or -inf) 0 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int vocab, int M, int N,
                                                     const int32_t* __restrict__ token_num, int32_t* __restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = row / N, r = row - b * N;
    if (r >= token_num[b]) {
        if (lane == 0) ids[row] = -1;
        return;
    }
    const float* src = logits + static_cast<int64_t>(row) * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    // 16 bytes per lane and four requests in flight where the row allows it; a lane meets its columns in rising order
    // either way, so the strict comparison keeps the lowest index of equal values
    int c0 = 0;
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        const int nq = vocab >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll 4
        for (int q = lane; q < nq; q += 64) {
            const float4 v = s4[q];
            if (v.x > best) { best = v.x; bi = 4 * q; }
            if (v.y > best) { best = v.y; bi = 4 * q + 1; }
            if (v.z > best) { best = v.z; bi = 4 * q + 2; }
            if (v.w > best) { best = v.w; bi = 4 * q + 3; }
        }
        c0 = nq * 4;
    }
    for (int c = c0 + lane; c < vocab; c += 64) {
        const float v = src[c];
        if (v > best) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // (a row with nothing above -inf, all NaN or all -inf, has no maximum: it reports index 0, never a value that
    // is not a vocabulary index)
    if (lane == 0) ids[row] = bi == 0x7fffffff ? 0 : bi;
}

// ---- arg-max from the vocabulary projection's own (value, column) pairs (one per 64-column strip): one wave per row,
// same tie rule and same row mask as argmax_kernel
__global__ __launch_bounds__(256) void argmax_pairs_kernel(const uint2* __restrict__ pairs, int ld, int M, int N,
                                                           const int32_t* __restrict__ token_num, int32_t* __restrict__ ids) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = row / N, r = row - b * N;
    if (r >= token_num[b]) {
        if (lane == 0) ids[row] = -1;
        return;
    }
    const uint2* src = pairs + static_cast<int64_t>(row) * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < ld; c += 64) {  // (strips in rising column order: the strict comparison keeps the lowest)
        const uint2 q = src[c];
        const float v = __uint_as_float(q.x);
        if (v > best) { best = v; bi = static_cast<int>(q.y); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) ids[row] = bi == 0x7fffffff ? 0 : bi;  // (see argmax_kernel)
}

// ---- launchers ----------------------------------------------------------------------------------------------------
#define DEC_CHECK_LAUNCH(who)                                                                                         \
    do {                                                                                                              \
        const hipError_t e__ = hipGetLastError();                                                                     \
        if (e__ != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string(who " launch: ") + hipGetErrorString(e__)); \
    } while (0)

int32_t launch_pred_split(asrfe_ctx* ctx, const float* enc, const int32_t* lens, int B, int Tp, uint16_t* hi, uint16_t* lo,
                          hipStream_t s) {
    const int M = B * Tp;
    if (M <= 0) return ASRFE_OK;
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16)
        hipLaunchKernelGGL(pred_split_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s, enc, lens, M, Tp, hi, lo);
    else
        hipLaunchKernelGGL(pred_split_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s, enc, lens, M, Tp, hi, lo);
    DEC_CHECK_LAUNCH("pred_split");
    return ASRFE_OK;
}

int32_t launch_pred_alpha(asrfe_ctx* ctx, const float* conv, const float* w, float bias, float smooth, float noise,
                          const int32_t* lens, int B, int Tp, float* alphas, hipStream_t s) {
    const int M = B * Tp;
    if (M <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(pred_alpha_kernel, dim3((M + 3) / 4), dim3(256), 0, s, conv, w, bias, smooth, noise, lens, M, Tp, alphas);
    DEC_CHECK_LAUNCH("pred_alpha");
    return ASRFE_OK;
}

int32_t launch_cif(asrfe_ctx* ctx, const CifParams& p, int B, hipStream_t s) {
    if (B <= 0 || p.Tp <= 0 || p.n_max <= 0) return ASRFE_OK;
    if ((p.Tp + 1) * 4 > 64 * 1024) return set_err(ctx, ASRFE_ERR_SHAPE, "cif: more than 16383 frames per utterance");
    hipLaunchKernelGGL(cif_kernel, dim3(B), dim3(256), (p.Tp + 1) * sizeof(float), s, p);
    DEC_CHECK_LAUNCH("cif");
    return ASRFE_OK;
}

int32_t launch_cif_fire_frames(asrfe_ctx* ctx, const float* alphas, int B, int Tp, int n_max, float thr, int32_t* fires,
                               hipStream_t s) {
    if (B <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(cif_fire_frames_kernel, dim3((B + 3) / 4), dim3(256), 0, s, alphas, B, Tp, n_max, thr, fires);
    DEC_CHECK_LAUNCH("cif_fire_frames");
    return ASRFE_OK;
}

int32_t launch_ln_hidden(asrfe_ctx* ctx, const uint16_t* in, uint16_t* out, int M, int width, float eps, hipStream_t s) {
    if (M <= 0) return ASRFE_OK;
    if (width % 512 != 0 || width < 512 || width > 4096)
        return set_err(ctx, ASRFE_ERR_SHAPE, "decoder: linear_units must be a multiple of 512 up to 4096");
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    const dim3 grid((M + 3) / 4), block(256);
#define LNH(NV)                                                                                          \
    do {                                                                                                 \
        if (f16) hipLaunchKernelGGL((ln_hidden_kernel<NV, true>), grid, block, 0, s, in, out, M, width, eps);  \
        else hipLaunchKernelGGL((ln_hidden_kernel<NV, false>), grid, block, 0, s, in, out, M, width, eps);     \
    } while (0)
    switch (width / 512) {
        case 1: LNH(1); break;
        case 2: LNH(2); break;
        case 3: LNH(3); break;
        case 4: LNH(4); break;
        case 5: LNH(5); break;
        case 6: LNH(6); break;
        case 7: LNH(7); break;
        default: LNH(8); break;
    }
#undef LNH
    DEC_CHECK_LAUNCH("ln_hidden");
    return ASRFE_OK;
}

int32_t launch_dec_fsmn(asrfe_ctx* ctx, const DecFsmnParams& p, int B, hipStream_t s) {
    if (B <= 0 || p.N <= 0) return ASRFE_OK;
    const dim3 grid((p.N + kDecFsmnRows - 1) / kDecFsmnRows, B);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(dec_fsmn_kernel<true>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(dec_fsmn_kernel<false>, grid, dim3(256), 0, s, p);
    DEC_CHECK_LAUNCH("dec_fsmn");
    return ASRFE_OK;
}

int32_t launch_argmax(asrfe_ctx* ctx, const float* logits, int ld, int vocab, int B, int N, const int32_t* token_num,
                      int32_t* ids, hipStream_t s) {
    const int M = B * N;
    if (M <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(argmax_kernel, dim3((M + 3) / 4), dim3(256), 0, s, logits, ld, vocab, M, N, token_num, ids);
    DEC_CHECK_LAUNCH("argmax");
    return ASRFE_OK;
}

int32_t launch_argmax_pairs(asrfe_ctx* ctx, const void* pairs, int ld, int B, int N, const int32_t* token_num, int32_t* ids,
                            hipStream_t s) {
    const int M = B * N;
    if (M <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(argmax_pairs_kernel, dim3((M + 3) / 4), dim3(256), 0, s, static_cast<const uint2*>(pairs), ld, M, N,
                       token_num, ids);
    DEC_CHECK_LAUNCH("argmax_pairs");
    return ASRFE_OK;
}

}  // namespace asrfe
