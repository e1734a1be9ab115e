// bf16 MFMA GEMM for the encoder linears on gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  (+ epilogue)
//
//   A : activations, bf16 [M][lda], K contiguous;  W : torch Linear weight, bf16 [N][ldw], K contiguous.
//   Tile 256 (m) x 256 (n) x 64 (k), 512 threads = 8 waves laid out 2 (m) x 4 (n); each wave owns a
//   128 (m) x 64 (n) piece as 4 x 2 tiles of v_mfma_f32_32x32x16_bf16.
//   Both operand tiles are staged global -> LDS with 16-byte LDS-DMA loads (no VGPR round trip) into
//   two buffers, so the loads of K-tile t+1 fly while K-tile t is multiplied.  An LDS-DMA wave
//   instruction writes 1 KiB linearly, so the bank swizzle is applied to the per-lane SOURCE address
//   and undone on the fragment read: 16-byte chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7),
//   which makes every ds_read_b128 lane group hit 16 distinct 16-byte slots.
//   The MFMA takes the W fragment as its A operand and the activation fragment as its B operand, so
//   the accumulator has n in registers (4 consecutive n per register quad) and m on lanes: the
//   epilogue then moves 16 bytes per lane (fp32 directly, bf16 after one v_permlane32_swap).
//   Workgroup ids are remapped so that the 8 XCDs each walk a contiguous range of tiles (their L2s
//   are private): the tiles that share an activation panel run on the same XCD at about the same time.
#include "common.h"
#include "elem.h"
#include "gemm.h"

namespace asrfe {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kGemmThreads = 512;
constexpr int kTileBytes = BM * BK * 2;      // 32 KiB per operand tile
constexpr int kBufBytes = 2 * kTileBytes;    // A tile + W tile
constexpr int kGemmLds = 2 * kBufBytes;      // 128 KiB

#define GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),       \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <int EPI, bool F16>
__global__ __launch_bounds__(kGemmThreads) void gemm_bf16_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware bijective remap of the workgroup id
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tm = logical / p.n_tiles_n, tn = logical - tm * p.n_tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // staging: thread covers 16-byte chunk (tid & 7) of LDS row i * 64 + (tid >> 3), i = 0..3
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);  // source chunk for that LDS slot
    const uint16_t* a_src[4];
    const uint16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + i * 64 + srow;
        ra = ra < p.M ? ra : p.M - 1;
        a_src[i] = p.A + static_cast<int64_t>(ra) * p.lda + schunk * 8;
        w_src[i] = p.W + static_cast<int64_t>(n0 + i * 64 + srow) * p.ldw + schunk * 8;
    }
    const int wave_slot = wave * 1024;  // byte offset of this wave's 1 KiB piece inside an 8 KiB pass

    auto stage = [&](int kt, int buf) {
        char* ab = smem + buf * kBufBytes + wave_slot;
        char* wb = ab + kTileBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            GLDS16(a_src[i] + kt * BK, ab + i * 8192);
            GLDS16(w_src[i] + kt * BK, wb + i * 8192);
        }
    };

    // fragment read offsets
    const int l31 = lane & 31, h = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) coff[s] = ((2 * s + h) ^ swz) << 4;
    const int a_row_off = (wm * 128 + l31) * 128;               // + ms * 32 * 128
    const int w_row_off = kTileBytes + (wn * 64 + l31) * 128;   // + ns * 32 * 128

    f32x16 acc[2][4];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ns][ms][e] = 0.0f;

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // K-tile kt has landed (the barrier's fence drains the LDS-DMA loads); tile kt-1 is consumed
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * kBufBytes;
        // fragments of k-step s + 1 are read while the MFMAs of k-step s run
        u16x8 af[2][4], wf[2][2];
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
            af[0][ms] = *reinterpret_cast<const u16x8*>(base + a_row_off + ms * 4096 + coff[0]);
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
            wf[0][ns] = *reinterpret_cast<const u16x8*>(base + w_row_off + ns * 4096 + coff[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
                    af[(s + 1) & 1][ms] = *reinterpret_cast<const u16x8*>(base + a_row_off + ms * 4096 + coff[s + 1]);
#pragma unroll
                for (int ns = 0; ns < 2; ++ns)
                    wf[(s + 1) & 1][ns] = *reinterpret_cast<const u16x8*>(base + w_row_off + ns * 4096 + coff[s + 1]);
            }
#pragma unroll
            for (int ns = 0; ns < 2; ++ns)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
                    acc[ns][ms] = mfma32<F16>(wf[s & 1][ns], af[s & 1][ms], acc[ns][ms]);
        }
    }

    // ---- epilogue: lane owns row m, register quad g owns n = nb + 8 g + 4 h + {0..3} ----------
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
        const int nb = n0 + wn * 64 + ns * 32;
        float bv[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 t = *reinterpret_cast<const float4*>(p.bias + nb + 8 * g + 4 * h);
            bv[4 * g] = t.x; bv[4 * g + 1] = t.y; bv[4 * g + 2] = t.z; bv[4 * g + 3] = t.w;
        }
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            const int m = m0 + wm * 128 + ms * 32 + l31;
            const bool ok = m < p.M;
            const f32x16 a = acc[ns][ms];
            if constexpr (EPI == EPI_F32_ADD) {
                float* orow = reinterpret_cast<float*>(p.out) + static_cast<int64_t>(m) * p.ldo;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + 8 * g + 4 * h;
                    float4 v = make_float4(a[4 * g] + bv[4 * g], a[4 * g + 1] + bv[4 * g + 1], a[4 * g + 2] + bv[4 * g + 2],
                                           a[4 * g + 3] + bv[4 * g + 3]);
                    if (ok) {
                        if (p.add1) {
                            const float4 t = *reinterpret_cast<const float4*>(p.add1 + static_cast<int64_t>(m) * p.ld1 + n);
                            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                        }
                        if (p.add2) {
                            const float4 t = *reinterpret_cast<const float4*>(p.add2 + static_cast<int64_t>(m) * p.ld2 + n);
                            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                        }
                        *reinterpret_cast<float4*>(orow + n) = v;
                    }
                }
            } else {
                uint16_t* orow = reinterpret_cast<uint16_t*>(p.out) + static_cast<int64_t>(m) * p.ldo;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = a[8 * gp + e] + bv[8 * gp + e];
                    if constexpr (EPI == EPI_BF16_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    uint32_t lo0 = pack_elem2<F16>(v[0], v[1]), lo1 = pack_elem2<F16>(v[2], v[3]);
                    uint32_t hi0 = pack_elem2<F16>(v[4], v[5]), hi1 = pack_elem2<F16>(v[6], v[7]);
                    // lanes 32..63 of lo swap with lanes 0..31 of hi: h = 0 ends with n 0..7, h = 1 with n 8..15
                    auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                    if (ok)
                        *reinterpret_cast<uint4*>(orow + nb + 16 * gp + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
        }
    }
}

int32_t launch_gemm(asrfe_ctx* ctx, const GemmParams& pin, int epi, hipStream_t s) {
    GemmParams p = pin;
    if (p.M <= 0) return ASRFE_OK;
    if (p.N % BN != 0 || p.K % BK != 0 || p.K <= 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || p.ldo % 8 != 0)
        return set_err(ctx, ASRFE_ERR_SHAPE, "gemm: N must be a multiple of 256, K of 64, leading dims of 8");
    p.n_tiles_m = (p.M + BM - 1) / BM;
    p.n_tiles_n = p.N / BN;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[6] = {reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, false>),
                              reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, false>),
                              reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, false>),
                              reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, true>),
                              reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, true>),
                              reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, true>)};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds) != hipSuccess)
                return set_err(ctx, ASRFE_ERR_HIP, "gemm: cannot raise the dynamic LDS limit to 128 KiB");
        attr_done = true;
    }
    dim3 grid(p.n_tiles_m * p.n_tiles_n), block(kGemmThreads);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
#define LAUNCH_GEMM(E)                                                                        \
    do {                                                                                      \
        if (f16) hipLaunchKernelGGL((gemm_bf16_kernel<E, true>), grid, block, kGemmLds, s, p); \
        else hipLaunchKernelGGL((gemm_bf16_kernel<E, false>), grid, block, kGemmLds, s, p);    \
    } while (0)
    switch (epi) {
        case EPI_BF16: LAUNCH_GEMM(EPI_BF16); break;
        case EPI_BF16_RELU: LAUNCH_GEMM(EPI_BF16_RELU); break;
        case EPI_F32_ADD: LAUNCH_GEMM(EPI_F32_ADD); break;
        default: return set_err(ctx, ASRFE_ERR_BAD_ARG, "gemm: bad epilogue");
    }
#undef LAUNCH_GEMM
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("gemm launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
