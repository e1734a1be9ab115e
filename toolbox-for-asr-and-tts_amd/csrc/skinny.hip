// GEMM for a handful of rows (one short utterance: tens to a few hundred rows) on gfx950.
//
// The throughput kernels give a wave 32 rows and ALL output columns, so a layer costs the same ~140 us for 50 rows as
// for 8000.  Here the work is cut the other way: a wave computes one 32-row x 32-column output tile over a slice of K,
// with both operands read straight from global memory (the activations are tiny and sit in L2, every weight element
// is needed by exactly one column tile), every k-step of both operands requested before the first MFMA: one memory
// round trip per launch instead of a pipelined stream.  N / 32 column tiles (x row-tile groups) workgroups spread a
// layer's few MB of weights over the whole chip.
//
//   A wave's K slice is 256 deep (16 k-steps: 128 registers of operands, all in flight at once); the KS = K / 256 waves
//   that share an output tile sum their partial tiles through LDS in a fixed order.  K = 512 packs two row tiles (RT)
//   into a workgroup, deeper products one.
// out = epilogue(A W^T + bias): 16-bit, 16-bit with ReLU, or f32 with optional 16-bit and f32 addends (which may
// alias out).  Rows at or past M are computed from row M - 1 and not stored.
#include "common.h"
#include "elem.h"
#include "gemm.h"

namespace asrfe {

constexpr int kSkinnySteps = 16;  // k-steps (of 16) per wave

template <bool F16, int EPI, int KS, int RT>
__global__ __launch_bounds__(64 * KS * RT) void skinny_gemm_kernel(const GemmParams p) {
    saturate_fp16_conversions<F16>();
    __shared__ float red[RT * KS * 16 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int rtl = wave / KS, ks = wave - rtl * KS;
    const int row0 = (blockIdx.y * RT + rtl) * 32;
    const int n0 = blockIdx.x * 32;
    const bool live = row0 < p.M;  // (a row tile past M still meets the barrier below)
    const int row = min(row0 + l31, p.M - 1);
    const uint16_t* wp = p.W + static_cast<int64_t>(n0 + l31) * p.ldw + ks * (16 * kSkinnySteps) + 8 * h;
    const uint16_t* ap = p.A + static_cast<int64_t>(row) * p.lda + ks * (16 * kSkinnySteps) + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    if (live) {
        u16x8 wf[kSkinnySteps], af[kSkinnySteps];
#pragma unroll
        for (int s = 0; s < kSkinnySteps; ++s) wf[s] = *reinterpret_cast<const u16x8*>(wp + 16 * s);
#pragma unroll
        for (int s = 0; s < kSkinnySteps; ++s) af[s] = *reinterpret_cast<const u16x8*>(ap + 16 * s);
        // every fragment is pinned to a register here, so all the loads above are in flight before the first MFMA
        // waits for one (left alone, the compiler keeps a few registers' worth in flight and the launch becomes a
        // chain of memory round trips)
#pragma unroll
        for (int s = 0; s < kSkinnySteps; ++s) asm volatile("" : "+v"(wf[s]), "+v"(af[s]));
#pragma unroll
        for (int s = 0; s < kSkinnySteps; ++s) acc = mfma32<F16>(wf[s], af[s], acc);
    }

    // register 4 g + e of the tile is column n0 + 8 g + 4 h + e of row row0 + l31
    auto finish = [&](int g, float v0, float v1, float v2, float v3) {
        const int m = row0 + l31, c = n0 + 8 * g + 4 * h;
        if (m >= p.M) return;
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + c);
        v0 += b4.x; v1 += b4.y; v2 += b4.z; v3 += b4.w;
        if constexpr (EPI == EPI_F32_ADD) {
            if (p.add16) {
                const uint2 t = *reinterpret_cast<const uint2*>(p.add16 + static_cast<int64_t>(m) * p.ld16 + c);
                float f0, f1, f2, f3;
                unpack_elem2<F16>(t.x, f0, f1);
                unpack_elem2<F16>(t.y, f2, f3);
                v0 += f0; v1 += f1; v2 += f2; v3 += f3;
            }
            if (p.add32) {
                const float4 t = *reinterpret_cast<const float4*>(p.add32 + static_cast<int64_t>(m) * p.ld32 + c);
                v0 += t.x; v1 += t.y; v2 += t.z; v3 += t.w;
            }
            *reinterpret_cast<float4*>(static_cast<float*>(p.out) + static_cast<int64_t>(m) * p.ldo + c) = make_float4(v0, v1, v2, v3);
        } else {
            if constexpr (EPI == EPI_BF16_RELU) {
                v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
            }
            *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.out) + static_cast<int64_t>(m) * p.ldo + c) =
                make_uint2(pack_elem2<F16>(v0, v1), pack_elem2<F16>(v2, v3));
        }
    };
    // partial tiles meet in LDS ([row tile][slice][register][lane]: conflict-free), then wave ks of a row tile finishes
    // the register quads g = ks, ks + KS, ... summed over the slices in a fixed order (no dependence on timing)
    float* rd = red + rtl * (KS * 16 * 64);
#pragma unroll
    for (int e = 0; e < 16; ++e) rd[(ks * 16 + e) * 64 + lane] = acc[e];
    __syncthreads();
    if (!live) return;
    for (int g = ks; g < 4; g += KS) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < KS; ++k) sum += rd[(k * 16 + 4 * g + e) * 64 + lane];
            v[e] = sum;
        }
        finish(g, v[0], v[1], v[2], v[3]);
    }
}

// K a multiple of 512 up to 3072 (12 slices: three waves per SIMD still leave each wave its 128 operand registers)
int32_t launch_skinny_gemm(asrfe_ctx* ctx, const GemmParams& p, int epi, hipStream_t s) {
    if (p.M <= 0) return ASRFE_OK;
    if (p.N % 32 != 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || p.ldo % 4 != 0 || !p.A || !p.W || !p.bias || !p.out)
        return set_err(ctx, ASRFE_ERR_SHAPE, "skinny gemm: N must be a multiple of 32, operands non-null and 16-byte aligned rows");
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    const int rts = (p.M + 31) / 32;
#define SK_LAUNCH(E, KSV, RTV)                                                                                   \
    do {                                                                                                         \
        const dim3 grid(p.N / 32, (rts + (RTV) - 1) / (RTV)), block(64 * (KSV) * (RTV));                          \
        if (f16) hipLaunchKernelGGL((skinny_gemm_kernel<true, E, KSV, RTV>), grid, block, 0, s, p);               \
        else hipLaunchKernelGGL((skinny_gemm_kernel<false, E, KSV, RTV>), grid, block, 0, s, p);                  \
    } while (0)
#define SK_EPI(KSV, RTV)                                                             \
    do {                                                                             \
        if (epi == EPI_BF16) SK_LAUNCH(EPI_BF16, KSV, RTV);                          \
        else if (epi == EPI_BF16_RELU) SK_LAUNCH(EPI_BF16_RELU, KSV, RTV);           \
        else if (epi == EPI_F32_ADD) SK_LAUNCH(EPI_F32_ADD, KSV, RTV);               \
        else return set_err(ctx, ASRFE_ERR_BAD_ARG, "skinny gemm: bad epilogue");    \
    } while (0)
    switch (p.K) {
        case 512: SK_EPI(2, 2); break;
        case 1024: SK_EPI(4, 1); break;
        case 1536: SK_EPI(6, 1); break;
        case 2048: SK_EPI(8, 1); break;
        case 2560: SK_EPI(10, 1); break;
        case 3072: SK_EPI(12, 1); break;
        default: return set_err(ctx, ASRFE_ERR_SHAPE, "skinny gemm: K must be a multiple of 512 up to 3072");
    }
#undef SK_EPI
#undef SK_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("skinny gemm launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
